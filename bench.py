#!/usr/bin/env python
"""bench.py -- BASELINE.json configs[1]: standalone BN254 G1 MSM 2^20 + Fr NTT 2^20 on MI355X.

A "step" is one pass of the hot path over one column: KZG-commit a 2^20-row column
(`commit_lagrange` = one 2^20 MSM over g_lagrange) and transform it (`lagrange_to_coeff` = one 2^20
NTT).  Inputs are resident in HBM before the timed region starts.  The steps walk a ROTATING set of
32 distinct columns to commit and 32 distinct columns to transform (2 GiB, eight times the 256 MiB
Infinity Cache), so no step finds its input in a cache the previous step filled.  Columns are
submitted the way a prover phase submits them: a batch of commitments (pipelined on the device),
then the batch of transforms (zk_ntt_batch).

`python bench.py --gpus N` launches itself as N ranks (torch.distributed.run, one rank per GPU,
backend nccl = RCCL) when it was not started under a launcher already.  Multi-GPU: the
prover shards by column -- rank r commits / transforms its own columns, no data-path collective;
the only exchange is the all-gather of the 64-byte commitments that a transcript round needs, done
once per commitment batch (`all_gather_into_tensor`), as the prover does per phase.  Weak scaling.

Prints ONE JSON line (rank 0).  `value` = scalars committed per second over all ranks (Mscalar/s)
with the step's NTT included in the time.  On one GPU the line also carries
  * `roofline` (dominant kernel, k_msm_buckets) and `rooflines` (every kernel class of the path),
  * `cpu_baseline` (the C oracle on the host cores),
  * `proof`: BASELINE's headline metric -- full-proof wall-clock of the SuperCircuit-shape circuit
    (k = 20) and of the Keccak-shape circuit (k = 18), each verified by the oracle's pairing verifier
    (synthetic-shape: the reference's witnesses need its Rust + Go toolchain).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 20
N = 1 << K
NCOL = 32                  # rotating set: 32 x 32 MiB committed + 32 x 32 MiB transformed (2 GiB, eight times the Infinity Cache)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MULPEAK_G = 169.0          # measured 9x29-bit Montgomery products/s (G) of the library's own product routine (tools/ubench.hip)
MAD_PEAK_T = 30.4          # measured v_mad_u64_u32 lane-ops/s (T), tools/ubench.hip: the hardware-side bound
MADS_PER_PRODUCT = 162     # v_mad_u64_u32 per 9 x 29-bit Montgomery product (81 operand + 81 reduction, counted in the ISA)
MADS_PER_MIXED_ADD = 1476  # per XYZZ mixed addition (csrc/ec29.hip.hpp madd29): 6 products x 162 + 2 squares x 126 + one two-product pass (243) + 9 (k p test)
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def relaunch_under_launcher(args) -> int:
    """`--gpus N` without a launcher: start N ranks of this script (one per GPU) and relay their output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_proof:
        cmd.append("--no-proof")
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proof", action="store_true", help="skip the full-proof section (N = 1 only)")
    ap.add_argument("--proof-worker", default="", help=argparse.SUPPRESS)      # internal: run ONE proof shape in this process and print its record
    ap.add_argument("--batch", type=int, default=32, help="columns submitted per commit_batch call (pipelined on the device; a prover phase commits tens to a thousand)")
    args = ap.parse_args()

    if args.proof_worker:
        print(json.dumps(proof_worker(args.proof_worker)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_launcher(args))

    import numpy as np
    import torch  # device plumbing + torch.distributed only

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    shared_gpu = world > ndev           # fewer GPUs than ranks (single-GPU test box): ranks share devices, exchange over gloo
    local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import zkevm_circuits_amd as z

    def fr_mont(v: int) -> np.ndarray:
        x = (v << 256) % R_MOD
        return np.array([(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)

    def synth_column(seed: int) -> np.ndarray:
        """n canonical Montgomery-form Fr values (252-bit uniform: always < r): dense scalars, every window occupied"""
        rng = np.random.default_rng(seed)
        a = rng.integers(0, 1 << 63, size=(N, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(N, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        bits = int(os.environ.get("ZK_BENCH_SCALAR_BITS", "0"))      # measurement knob: witness-like small values (Montgomery images of integers < 2^bits)
        if 0 < bits <= 60:
            vals = rng.integers(0, 1 << bits, size=N, dtype=np.uint64)
            mont = [(int(v) << 256) % R_MOD for v in vals]           # Montgomery images, big-int arithmetic (a second or two for 2^20)
            return np.array([[(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)] for x in mont], dtype=np.uint64)
        return a

    stream = torch.cuda.current_stream().cuda_stream
    ctx = z.Context(local_rank, stream=stream if stream else None)
    srs = ctx.srs_setup_with_s(K, fr_mont(0xC0FFEE))
    first_column = synth_column(1000 * (rank + 1))
    d_cols = [ctx.to_device(first_column if i == 0 else synth_column(1000 * (rank + 1) + i)) for i in range(NCOL)]       # committed (read-only)
    d_work = [ctx.to_device(synth_column(5000 * (rank + 1) + i)) for i in range(NCOL)]                                   # transformed in place
    gather = None
    xdev = "cpu" if shared_gpu else "cuda"
    com_t = torch.zeros(64 * args.batch, dtype=torch.uint8, device=xdev)
    if world > 1:
        gather = torch.zeros(64 * args.batch * world, dtype=torch.uint8, device=xdev)
    cursor = [0]

    def run_steps(count):
        """`count` steps = `count` columns: the prover commits the columns of a phase as a batch
        (halo2: commit_lagrange over every advice column), so consecutive MSMs are pipelined."""
        done = 0
        while done < count:
            b = min(args.batch, count - done)
            ids = [(cursor[0] + j) % NCOL for j in range(b)]
            cursor[0] += b
            coms = ctx.commit_batch(srs, [d_cols[i].ptr for i in ids], N, lagrange=True)    # b x MSM 2^20, b distinct columns
            ctx.ntt_batch([d_work[i] for i in ids], K, inverse=True)                       # b x NTT 2^20 (lagrange_to_coeff), b distinct buffers, as the prover
                                                                                          # transforms the columns of a round (zk_ntt_batch: four columns share a launch)
            if world > 1:
                # one exchange per commitment round, as in the prover: every rank needs every
                # commitment of the batch (64 B each) before the next transcript challenge
                com_t[:64 * b].copy_(torch.from_numpy(np.ascontiguousarray(coms).view(np.uint8).reshape(-1)))
                dist.all_gather_into_tensor(gather, com_t)
            done += b

    def device_sync():
        # the library's streams belong to the HIP runtime it links (/opt/rocm), torch's to the one torch bundles:
        # torch.cuda.synchronize() alone would not wait for the transforms the last batch left in flight
        ctx.sync()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    ctx.prof_reset()
    ctx.prof_enable(2)           # HIP events around the roofline kernels only (bucket accumulation, the two NTT passes)
    t0 = time.perf_counter()
    run_steps(args.steps)
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof_timed = {name: ctx.prof_get(name) for name in ctx.prof_names()}
    # the other kernel groups (sort, combine, reduction) are timed in a pass of their own, outside the timed region
    ctx.prof_reset()
    ctx.prof_enable(True)
    run_steps(min(args.steps, 2 * args.batch))
    device_sync()
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {name: ctx.prof_get(name) for name in ctx.prof_names()}
    prof.update(prof_timed)      # the roofline kernels keep their timed-region figures
    # latency of ONE commitment issued alone (nothing to hide its bucket reduction under), wall clock around a synchronous call
    lone = []
    for i in range(6):
        t1 = time.perf_counter()
        ctx.commit(srs, d_cols[i % NCOL], N, lagrange=True)
        lone.append(time.perf_counter() - t1)
    lone_ms = sorted(lone)[len(lone) // 2] * 1e3
    # N > 1 on real GPUs: the sharded proving session of the SuperCircuit shape (BASELINE config 4 is quoted on 8 GPUs) and of
    # the recursion shape (config 5), every rank in a prover process of its own, exchanges through the library's RCCL
    # communicator.  All ranks take part; only rank 0 gets the records.  Failures and time-outs stay inside the section.
    sharded = None
    if world > 1 and not shared_gpu and not args.no_proof:
        for b_ in d_cols + d_work:
            b_.free()
        d_cols, d_work = [], []
        try:
            sharded = sharded_proof_section(dist, rank, world, local_rank)
        except Exception as e:           # the MSM / NTT line must survive whatever happens in here
            sharded = {"error": repr(e)} if rank == 0 else None
        dist.barrier()
    if rank == 0:
        def avg_ms(name):
            ms, cnt = prof.get(name, (0.0, 0))
            return ms / cnt if cnt else None

        def hbm_roof(kernel, alg_bytes, ms, note, products=None):
            """roofline record of one kernel class: algorithmic bytes per launch / measured launch time
            against the HBM peak (the metric's roof) plus, where given, the integer-ALU roof that binds"""
            if not ms:
                return None
            ach = alg_bytes / (ms * 1e-3) / 1e9
            rec = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                   "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes), "note": note}
            if products:
                gps = products / (ms * 1e-3) / 1e9
                rec["alu"] = {"unit": "G Montgomery-product equivalents/s (162 multiply-adds each)", "achieved": round(gps, 1), "peak_own_routine": MULPEAK_G, "frac_own_routine": round(gps / MULPEAK_G, 3),
                              "peak_v_mad_u64_u32": round(MAD_PEAK_T * 1e3 / MADS_PER_PRODUCT, 1), "frac_v_mad_u64_u32": round(gps / (MAD_PEAK_T * 1e3 / MADS_PER_PRODUCT), 3)}
            return rec

        plan = ctx.msm_plan(srs, N) if hasattr(ctx, "msm_plan") else {"c": 16, "windows": 16}
        windows = plan["windows"]
        bucket_ms = avg_ms("msm_buckets")
        sort_ms, comb_ms, red_ms = avg_ms("msm_sort") or 0.0, avg_ms("msm_combine") or 0.0, avg_ms("msm_reduce") or 0.0
        msm_pipelined_ms = sort_ms + (bucket_ms or 0.0) + comb_ms            # the reduction runs on the side stream under the next MSM
        msm_lone_ms = lone_ms                                                # what one MSM alone costs (measured above): nothing hides its reduction
        ntt_ms = (prof.get("ntt_pass", (0, 0))[0] + prof.get("ntt_last", (0, 0))[0]) / max(args.steps, 1)
        traffic, tsrc = None, None
        for tname in ("traffic_r03.json", "traffic_r02.json"):           # the newest committed rocprofv3 --pmc pass of this command
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                try:
                    traffic, tsrc = json.load(open(tpath)).get("msm_buckets_bytes_per_launch"), tname
                except Exception:
                    traffic = None
                if traffic:
                    break
        main_roof = hbm_roof("k_msm_buckets", 96.0 * N, bucket_ms,
                             "algorithmic bytes = 96 B (32 B scalar + 64 B affine base) x 2^20 (SURVEY 8d); integer-ALU bound: one mixed XYZZ addition "
                             f"({MADS_PER_MIXED_ADD} multiply-adds = {MADS_PER_MIXED_ADD / MADS_PER_PRODUCT:.2f} Montgomery-product equivalents) per (scalar, window), {windows} windows",
                             products=MADS_PER_MIXED_ADD / MADS_PER_PRODUCT * N * windows)
        if main_roof:
            main_roof["traffic"] = traffic            # PMC FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per launch, from the committed rocprofv3 pass (profiles/); null until measured this round
            main_roof["traffic_source"] = f"profiles/{tsrc} (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of this command, per launch, as reported: the guide's x2 FETCH correction is for coalesced streaming reads and these are 64-byte gathers -- profiles/r03_pmc_traffic.md gives both)" if traffic else None
        rooflines = [r for r in (
            main_roof,
            hbm_roof("k_ntt_pass + k_ntt_last (one 2^20 transform)", 64.0 * N, ntt_ms,
                     "algorithmic bytes = 64 B x 2^20 (read once, write once); VALU-issue bound: 10.5 M Montgomery products per transform", products=N * K / 2.0),
        ) if r]
        out = {
            "metric": "MSM Mscalar/s (step = KZG commit of one 2^20 column: MSM 2^20 + NTT 2^20)",
            "value": round(world * N * args.steps / elapsed / 1e6, 3),
            "unit": "Mscalar/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32x8 limbs (254-bit modular integer, Montgomery)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: BN254 G1 MSM 2^20 + Fr NTT 2^20 per step, 1 column per step per GPU", "k": K,
                       "parallelism": f"column-sharded x{world}, all_gather(64 B commitment per column) once per commit batch, backend {'gloo (ranks share a GPU)' if shared_gpu else 'nccl (RCCL)'}" if world > 1 else "single GPU",
                       "columns_per_commit_batch": args.batch, "rotating_columns": f"{NCOL} committed + {NCOL} transformed, 32 MiB each ({2 * NCOL * 32 >> 10} GiB working set)",
                       "msm_window_bits": plan.get("c"), "msm_windows": windows},
            "roofline": main_roof,
            "rooflines": rooflines,
            "extra": {
                "msm_pipelined_ms": round(msm_pipelined_ms, 4),
                "msm_pipelined_mscalar_per_s": round(N / (msm_pipelined_ms * 1e-3) / 1e6, 2) if msm_pipelined_ms else None,
                "msm_lone_ms": round(msm_lone_ms, 4),
                "msm_reduce_ms_on_side_stream": round(red_ms, 4),
                "ntt_only_ms": round(ntt_ms, 4),
                "ntt_gfieldop_per_s": round(1.5 * N * K / (ntt_ms * 1e-3) / 1e9, 2) if ntt_ms else None,
                "ntt_algorithmic_GBps": round(64.0 * N / (ntt_ms * 1e-3) / 1e9, 1) if ntt_ms else None,
                "kernel_avg_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items() if v[1]},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(srs, first_column)
        for b_ in d_cols + d_work:
            b_.free()
        srs.destroy()
        if world == 1 and not args.no_proof:
            out["proof"] = proof_section()
        if sharded is not None:
            out["proof_sharded"] = sharded
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()          # streams, events and the scratch arenas go before the interpreter tears HIP down


def cpu_baseline(srs, column):
    """Oracle (C restatement of halo2's best_multiexp + best_fft, OpenMP) on the host cores:
    one 2^20 MSM + one 2^20 NTT = exactly one bench step.  Reported, never the target."""
    from oracle import bn254, cref

    bases = srs.download_g_lagrange()
    threads = cref.num_threads()
    t0 = time.perf_counter()
    cref.best_multiexp(column, bases, threads)
    t1 = time.perf_counter()
    cref.best_fft(column, bn254.omega_for_k(K), K)
    t2 = time.perf_counter()
    return {
        "value": round(N / (t2 - t0) / 1e6, 4),
        "unit": "Mscalar/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one full step on the host: MSM 2^20 ({t1 - t0:.2f} s) + NTT 2^20 ({t2 - t1:.2f} s), C oracle (halo2 best_multiexp/best_fft restated), OpenMP {threads} threads",
        "msm_s": round(t1 - t0, 3),
        "ntt_s": round(t2 - t1, 3),
    }


PROOF_SHAPES = ("keccak_shape_k18", "recursion_shape_k22", "supercircuit_shape_k20", "keccak_shape_k16_cpu_vs_gpu", "evm_shape_k14_mock")
MOCK_SHAPES = {"evm_shape_k14_mock": ("build_large", (14, 53)),                           # BASELINE configs[0] stand-in: k = 14, 159 advice columns
               "supercircuit_shape_k20_mock": ("build_shape", (20, 1000, 150, 150, 100, 9))}      # only with ZK_BENCH_PROOFS=supercircuit_shape_k20_mock


def proof_section():
    """BASELINE's headline metric on one GPU: full-proof wall-clock of the SuperCircuit-shape circuit
    (config 4 stand-in, k = 20: 1000 advice / 150 fixed / 150 permutation columns, 100 lookups, degree 9),
    of the Keccak-shape circuit (config 3 stand-in, k = 18: 59 unusable rows, 13-rotation gates, degree 9)
    and of the recursion shape (config 5 stand-in), SHPLONK as at [REF circuit-benchmarks/src/super_circuit.rs:117-132];
    each proof is checked by the oracle's pairing verifier.  Every shape runs in a process of its own
    (`bench.py --proof-worker <shape>`: a prover process holding the library and nothing else.  This process has
    torch loaded for the launcher contract, i.e. torch's bundled HIP runtime next to the one the library links; with
    both in one process the advice-phase uploads measured 30 % slower -- 0.88 against 0.66 ms per 32 MiB column,
    tools/upload_order.py -- and a Rust / C prover has no torch in it)."""
    import subprocess

    out = {}
    only = os.environ.get("ZK_BENCH_PROOFS", "")          # measurement knob: comma-separated subset of the shapes
    for name in PROOF_SHAPES:
        if only and name not in only.split(","):
            continue
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--proof-worker", name], capture_output=True, text=True, timeout=900)
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not lines:
                out[name] = {"error": f"worker exited with {res.returncode}: {res.stderr[-400:]}"}
            else:
                out[name] = json.loads(lines[-1])
            if os.environ.get("ZK_PROVER_TRACE"):
                sys.stderr.write(res.stderr)
        except Exception as e:           # the MSM / NTT line must survive a failure of the proof section
            out[name] = {"error": repr(e)}
    return out


def sharded_proof_section(dist, rank, world, local_rank):
    """One prover process per rank (`--proof-worker <shape>` with the launcher's RANK / WORLD_SIZE / LOCAL_RANK in its
    environment): the ranks join the library's own RCCL communicator through a file (zkevm-circuits_amd/rendezvous.py) and
    run the sharded session -- commitments split by column (by points when there are fewer columns than ranks), quotient
    split by coset, witness columns uploaded by their owner and all-gathered device to device.  Returns the records on
    rank 0, None elsewhere."""
    import subprocess

    def host_bytes_free():
        """what this container may still take: the cgroup's limit if it has one, the machine's free memory otherwise"""
        free = None
        try:
            import psutil
            free = psutil.virtual_memory().available
        except Exception:
            pass
        try:
            lim = open("/sys/fs/cgroup/memory.max").read().strip()
            if lim != "max":
                cur = int(open("/sys/fs/cgroup/memory.current").read())
                free = min(free, int(lim) - cur) if free is not None else int(lim) - cur
        except Exception:
            pass
        return free

    out = {}
    need = {"supercircuit_shape_k20": 24 << 30, "recursion_shape_k22": 8 << 30}      # host bytes per rank (blob + witness + builder temporaries)
    for name, limit in (("supercircuit_shape_k20", 150), ("recursion_shape_k22", 120)):      # expected: 20-30 s and 15 s
        verdict = [None]
        if rank == 0:
            free = host_bytes_free()
            if free is not None and free < world * need[name]:
                verdict[0] = f"host memory: {free >> 30} GiB free, {world} ranks x {need[name] >> 30} GiB needed"
        dist.broadcast_object_list(verdict, src=0)               # one verdict for all ranks
        if verdict[0]:
            if rank == 0:
                out[name] = {"skipped": verdict[0]}
            continue
        env = dict(os.environ)
        env["ZK_COMM_ID_FILE"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"zkmi355_comm_{os.environ.get('MASTER_PORT', '29500')}_{os.getppid()}_{name}")
        env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = str(rank), str(world), str(local_rank)
        failed = False
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--proof-worker", name], capture_output=True, text=True, timeout=limit, env=env)
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            failed = res.returncode != 0
            if rank == 0:
                out[name] = json.loads(lines[-1]) if not failed and lines else {"error": f"rank 0 worker exited with {res.returncode}: {res.stderr[-400:]}"}
        except Exception as e:           # a time-out here usually means another rank failed and the collectives never completed
            failed = True
            if rank == 0:
                out[name] = {"error": repr(e)}
        if rank == 0:
            try:
                os.remove(env["ZK_COMM_ID_FILE"])
            except OSError:
                pass
        if failed:                       # every rank sees the failure (its own exit code or the common time-out): none starts the next shape
            break
    return out if rank == 0 else None


def cpu_vs_gpu_worker(k=16):
    """The proof-level CPU baseline, MEASURED (SURVEY 8d last row; the reference's `[Proof generation]` timer
    [REF circuit-benchmarks/src/super_circuit.rs:115-134] needs Rust): halo2's create_proof restated over arrays with the
    oracle's C primitives and OpenMP (oracle/cpu_prover.py: whole-extended-domain evaluate_h, as upstream's CPU prover), on the
    Keccak shape at k = 16 -- a quarter of the rows of BASELINE config 3, so that the run stays around ten to thirty seconds of
    host time -- beside the GPU session on the SAME circuit, witness, seed and vk.transcript_repr: the two proofs must be the
    same bytes.  Key generation (fixed / sigma forms, halo2's keygen_pk) and the SRS are outside both timings."""
    import numpy as np

    import bench_proof as bp
    import zkevm_circuits_amd as z
    from oracle import cpu_prover as cp, cref

    S = 0x5EC2E7
    ctx = z.Context(0)
    circ, blob, adv_m, inst_m, inst = bp.build_keccak_shape(ctx, k)
    npub = [int(np.flatnonzero(np.asarray(a).reshape(-1, 4).any(axis=1))[-1]) + 1 if np.asarray(a).any() else 0 for a in inst_m]
    inst_m = [np.ascontiguousarray(a[:m]) for a, m in zip(inst_m, npub)]
    pinned = []
    for a in adv_m:
        h = ctx.host_alloc(a.shape)
        h[:] = a
        pinned.append(h)
    srs = ctx.srs_setup_with_s(k, cref.fr_const(S))
    pk = ctx.pk_create(srs, blob)
    _, rep = pk.vk(circ.F + len(circ.perm_cols))
    repr_int = cref.from_mont(rep.reshape(1, 4))[0]
    gpu_times, gpu_proof = [], b""
    for _ in range(4):
        t0 = time.perf_counter()
        sess = ctx.proof_session(pk, inst_m, bytes(16), instance_slices=True)
        sess.set_multiopen(1)
        sess.advice_phase({i: c for i, c in enumerate(pinned)})
        gpu_proof = sess.finish()
        gpu_times.append(time.perf_counter() - t0)
    pk.destroy()
    srs.destroy()
    ctx.close()
    circ_h, adv_h, inst_h = bp.build_keccak_shape(None, k)          # same seed: the same circuit and witness, host side only
    key = cp.keygen(circ_h)
    srs_h = cp.Srs(k, S)
    stages = {}
    t0 = time.perf_counter()
    cpu_proof = cp.create_proof(circ_h, srs_h, adv_h, inst_h, repr_int, bytes(16), "shplonk", timings=stages, key=key)
    cpu_s = time.perf_counter() - t0
    threads = cref.num_threads()
    return {
        "metric": "synthetic-shape full proof wall-clock (s): restated CPU prover vs 1x MI355X, same circuit / witness / seed",
        "shape": f"Keccak shape at k = {k} ({circ.A} advice, {circ.F} fixed, {len(circ.perm_cols)} permutation columns, {len(circ.lookups)} lookups, degree {circ.degree()}, {circ.bf} blinding factors)",
        "cpu_baseline": {"value": round(cpu_s, 3), "unit": "s", "cores": threads, "kind": "port",
                         "sample": f"ONE full proof of the k = {k} Keccak shape: halo2 create_proof restated over arrays (oracle/cpu_prover.py), C primitives, OpenMP {threads} threads; "
                                   "keygen and SRS outside the timing; not the reference's Rust prover (no toolchain here)",
                         "stages_s": {name: round(v, 3) for name, v in stages.items()}},
        "value": round(min(gpu_times), 4), "unit": "s", "higher_is_better": False,
        "gpu_s": round(min(gpu_times), 4), "gpu_create_proof_s": [round(t, 4) for t in gpu_times],
        "speedup_vs_restated_cpu": round(cpu_s / min(gpu_times), 1),
        "same_proof_bytes": cpu_proof == gpu_proof, "proof_bytes": len(gpu_proof), "data": "synthetic-shape",
    }


def mock_worker(name):
    """BASELINE configs[0] (the reference's own CPU-runnable case): MockProver over the EVM sub-circuit at k = 14
    [REF circuit-benchmarks/src/evm_circuit.rs:44-60] -- here zk_mock_verify (dev::MockProver::verify_par restated for the
    device) over the same-size stand-in: the satisfied witness, then one cell changed (the path that lists failures)."""
    import numpy as np
    import bench_proof as bp
    import zkevm_circuits_amd as z
    from zkevm_circuits_amd import plonk

    ctx = z.Context(0)
    builder, args = MOCK_SHAPES[name]
    circ, blob, adv_m, inst_m, inst = getattr(bp, builder)(ctx, *args)
    srs = ctx.srs_setup_with_s(circ.k, np.frombuffer(plonk.fr_mont_bytes(0x5EC2E7), dtype=np.uint64).copy())
    pk = ctx.pk_create(srs, blob)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        got, total = ctx.mock_verify(pk, adv_m, inst_m)
        times.append(time.perf_counter() - t0)
        assert total == 0, got[:4]
    bad = [np.array(c, copy=True) for c in adv_m]
    row = int(np.flatnonzero(np.asarray(bad[2]).any(axis=1))[0])            # a cell a gate reads: first non-zero cell of advice column 2
    bad[2][row, 0] ^= np.uint64(1)
    t0 = time.perf_counter()
    got, total = ctx.mock_verify(pk, bad, inst_m)
    t_bad = time.perf_counter() - t0
    assert total >= 1
    rec = {"metric": "MockProver-style witness check wall-clock (s), zk_mock_verify on 1 x MI355X, witness columns uploaded from host memory inside the timed call", "value": round(min(times), 4),
           "unit": "s", "higher_is_better": False, "k": circ.k, "advice_columns": circ.A, "gate_polynomials": len(circ.gates), "lookups": len(circ.lookups),
           "permutation_columns": len(circ.perm_cols), "runs_s": [round(t, 4) for t in times],
           "one_cell_changed": {"wall_s": round(t_bad, 4), "failures": total, "first": list(got[0]) if got else None},
           "data": "synthetic circuit of the configuration's size class (the EVM circuit itself needs the Rust exporter)"}
    pk.destroy()
    srs.destroy()
    ctx.close()
    return rec


def proof_worker(name):
    """One proof shape, measured in this (fresh) process.  The quotient evaluator's roofline comes from one extra,
    profiled proof: bytes = what the launches really stream (counted by the library) over their time."""
    import bench_proof as bp
    import zkevm_circuits_amd as z

    if name == "keccak_shape_k16_cpu_vs_gpu":
        return cpu_vs_gpu_worker(16)
    if name in MOCK_SHAPES:
        return mock_worker(name)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    ctx = z.Context(int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0)
    hook = barrier = None
    if world > 1:
        from zkevm_circuits_amd import rendezvous

        rendezvous.comm_init_from_env(ctx)
        hook = lambda sess: sess.set_sharding_comm()
        barrier = lambda: rendezvous.comm_barrier(ctx, rank, world)
    # (builder, proofs per key, transcript): the recursion shape is BASELINE config 5's stand-in -- k = 22, 9 advice
    # columns, FOUR sequential proofs sharing one proving key, Poseidon transcript as gen_snark_shplonk uses
    # [REF prover/src/common/prover/recursion.rs:60-77], [REF aggregator/configs/bundle_circuit.config]
    build, repeat, tkind = {"keccak_shape_k18": (lambda: bp.build_keccak_shape(ctx, 18), 3, None),
                            "recursion_shape_k22": (lambda: bp.build_large(ctx, 22, 3), 4, 1),
                            "supercircuit_shape_k20": (lambda: bp.build_shape(ctx, 20, 1000, 150, 150, 100, 9), 4, None)}[name]
    t0 = time.perf_counter()
    circ, blob, adv_m, inst_m, inst = build()
    t_build = time.perf_counter() - t0
    rec = bp.proof_bench(ctx, circ.k, circ, blob, adv_m, inst_m, inst, shplonk=True, repeat=repeat, verify=True, pinned=True, t_build=t_build, transcript_kind=tkind,
                         profiled_extra=world == 1,  # timed proofs run without the profiling events; one extra proof feeds the quotient roofline
                         session_hook=hook, barrier=barrier, report=rank == 0, world=world)
    if world > 1:
        if rec is not None:
            rec["metric"] = f"synthetic-shape full proof wall-clock (s), sharded session on {world} x MI355X (in-library RCCL)"
            if tkind == 1:
                rec["transcript"] = "poseidon"
                rec["chain_of_4_proofs_s"] = round(sum(rec["create_proof_s"]), 4)
        ctx.close()
        return rec
    if tkind == 1:
        rec["transcript"] = "poseidon"
        rec["chain_of_4_proofs_s"] = round(sum(rec["create_proof_s"]), 4)
    prof = {nm: ctx.prof_get(nm) for nm in ctx.prof_names()}
    n = 1 << circ.k
    d, P, L = circ.degree(), len(circ.perm_cols), len(circ.lookups)
    C = (P + d - 3) // (d - 2) if P else 0
    # distinct (column, rotation) operands of the quotient program: the circuit's own queries, sigma, Z (x, wx, w^last x),
    # phi (x, wx) and m per lookup, l_0 / l_last / l_active / X
    reads = len(circ.advice_queries) + len(circ.fixed_queries) + len(circ.instance_queries) + P + (3 * C - 1 if C else 0) + 3 * L + 4
    cosets = 1 << (circ.extended_k() - circ.k)
    qbig = prof.get("quotient_coset", (0.0, 0))
    if qbig[1]:
        # The quotient is evaluated by degree class: a proof launches one program per (class, coset of that
        # class) instead of one per coset.  `achieved` = the bytes those launches really stream (the library counts
        # 32 B x rows x (distinct (column, rotation) operands + parked intermediates + 1 result) per launch) over their
        # time.  `vs_full_domain` = what evaluating every constraint on every coset would stream (halo2's evaluate_h)
        # over the same time: an EFFECTIVE rate, comparable across rounds, that may exceed the HBM peak.
        per_proof_ms = qbig[0]                       # the profiled proof
        streamed = ctx.prof_get_bytes("quotient_coset")
        full = 32.0 * n * (reads + 1) * cosets
        rec["roofline_quotient"] = {"kernel": "k_quotient_eval (all degree-class launches of one proof)", "bound": "hbm",
                                    "achieved": round(streamed / (per_proof_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(streamed / (per_proof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_proof": round(per_proof_ms, 2),
                                    "launches_per_proof": qbig[1], "algorithmic_bytes_per_proof": int(streamed),
                                    "distinct_column_rotation_reads": reads, "cosets": cosets,
                                    "vs_full_domain": {"bytes": int(full), "effective_GBps": round(full / (per_proof_ms * 1e-3) / 1e9, 1)}}
        try:            # counters of the committed rocprofv3 --pmc passes over the evaluator (profiles/r03_quotient_traffic.md): FETCH_SIZE + WRITE_SIZE, as reported
            qt = json.load(open(os.path.join(ROOT, "profiles", "traffic_r03.json"))).get("quotient", {})
            if name == "supercircuit_shape_k20" and "supercircuit_shape_proof" in qt:
                t_ = qt["supercircuit_shape_proof"]
                rec["roofline_quotient"]["traffic"] = t_["fetch_bytes_raw"] + t_["write_bytes"]
                rec["roofline_quotient"]["traffic_note"] = f"all {t_['launches']} launches of the kernel in one proof (the coset programs AND the theta-compression / permutation / linear-combination programs), rocprofv3 --pmc, profiles/r03_quotient_traffic.md"
            ql = qt.get("quot_loop")
            if ql:
                rec["roofline_quotient"]["counter_over_algorithmic_on_the_gate_loop"] = round((ql["fetch_bytes_raw"] + ql["write_bytes"]) / ql["algorithmic_bytes"], 3)
        except Exception:
            pass
    ctx.close()
    return rec


if __name__ == "__main__":
    main()
