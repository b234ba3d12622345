#!/usr/bin/env python
"""bench.py -- BASELINE.json configs[1]: standalone BN254 G1 MSM 2^20 + Fr NTT 2^20 on MI355X.

A "step" is one pass of the hot path over one column: KZG-commit a 2^20-row column
(`commit_lagrange` = one 2^20 MSM over g_lagrange) and transform it (`lagrange_to_coeff` = one 2^20
NTT).  Inputs (column, SRS) are resident in HBM before the timed region starts.

Multi-GPU: the prover shards by column -- rank r commits/transforms its own column,
no data-path collective; the only exchange is the all-gather of the 64-byte commitments that a
transcript round needs, done here once per commitment batch with RCCL
(`torch.distributed.all_gather_into_tensor`), as the prover does per phase.  Weak
scaling: per-GPU work is fixed.

Prints ONE JSON line (rank 0).  `value` = scalars committed per second over all ranks (Mscalar/s)
with the step's NTT included in the time; the MSM-only and NTT-only rates are under "extra".
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device plumbing + torch.distributed only)

K = 20
N = 1 << K
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MULPEAK_G = 169.0          # measured 9x29-bit Montgomery products/s (G), tools/ubench.hip
MSM_WINDOWS = 16           # c = 16 signed digits at n = 2^20


R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def fr_mont(v: int) -> np.ndarray:
    """Montgomery image (R = 2^256) of a small integer, as 4 x u64 limbs."""
    x = (v << 256) % R_MOD
    return np.array([(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def synth_column(seed: int) -> np.ndarray:
    """n canonical Montgomery-form Fr values (252-bit uniform: always < r)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(N, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(N, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="columns submitted per commit_batch call (pipelined on the device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    stream = torch.cuda.current_stream().cuda_stream
    ctx = z.Context(local_rank, stream=stream if stream else None)
    srs = ctx.srs_setup_with_s(K, fr_mont(0xC0FFEE))
    column = synth_column(1234 + rank)
    d_col = ctx.to_device(column)            # committed every step (read-only)
    d_work = ctx.to_device(column)           # transformed in place every step
    gather = None
    xdev = "cpu" if shared_gpu else "cuda"
    com_t = torch.zeros(64 * args.batch, dtype=torch.uint8, device=xdev)
    if world > 1:
        gather = torch.zeros(64 * args.batch * world, dtype=torch.uint8, device=xdev)

    def run_steps(count):
        """`count` steps = `count` columns: the prover commits the columns of a phase as a batch
        (halo2: commit_lagrange over every advice column), so consecutive MSMs are pipelined."""
        done = 0
        while done < count:
            b = min(args.batch, count - done)
            coms = ctx.commit_batch(srs, [d_col.ptr] * b, N, lagrange=True)   # b x MSM 2^20
            for _ in range(b):
                ctx.ntt(d_work, K, inverse=True)                               # b x NTT 2^20 (lagrange_to_coeff)
            if world > 1:
                # one exchange per commitment round, as in the prover: every rank needs every
                # commitment of the batch (64 B each) before the next transcript challenge
                com_t[:64 * b].copy_(torch.from_numpy(np.ascontiguousarray(coms).view(np.uint8).reshape(-1)))
                dist.all_gather_into_tensor(gather, com_t)
            done += b

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {name: ctx.prof_get(name) for name in ctx.prof_names()}
    if rank == 0:
        def avg_ms(name):
            ms, cnt = prof.get(name, (0.0, 0))
            return ms / cnt if cnt else None

        bucket_ms = avg_ms("msm_buckets")
        msm_ms = sum(avg_ms(x) or 0.0 for x in ("msm_sort", "msm_buckets", "msm_combine"))   # reduce runs on the side stream under the next MSM
        ntt_ms = (prof.get("ntt_pass", (0, 0))[0] + prof.get("ntt_last", (0, 0))[0]) / max(args.steps, 1)
        # roofline of the dominant kernel (bucket accumulation): algorithmic bytes per launch =
        # 96 B/unit (32 B scalar + 64 B affine base) x 2^20 units
        alg_bytes = 96.0 * N
        achieved = alg_bytes / (bucket_ms * 1e-3) / 1e9 if bucket_ms else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("msm_buckets_bytes_per_launch")
            except Exception:
                traffic = None
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
            "config": {"workload": "BASELINE configs[1]: BN254 G1 MSM 2^20 + Fr NTT 2^20 per step, 1 column per GPU", "k": K,
                       "parallelism": f"column-sharded x{world}, all_gather(64 B commitment per column) once per commit batch" if world > 1 else "single GPU",
                       "columns_per_commit_batch": args.batch},
            "roofline": {
                "kernel": "k_msm_buckets",
                "bound": "hbm",
                "achieved": round(achieved, 2) if achieved else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                "traffic": traffic,
                "avg_launch_ms": round(bucket_ms, 4) if bucket_ms else None,
                "traffic_GBps": round(traffic / (bucket_ms * 1e-3) / 1e9, 1) if (traffic and bucket_ms) else None,
                # the binding roof (SURVEY 8d asks for both): one mixed XYZZ addition per (scalar, window) = 10
                # Montgomery products, against the measured product peak of tools/ubench.hip
                "alu": {"unit": "G Montgomery products/s", "peak": MULPEAK_G,
                        "achieved": round(10.0 * N * MSM_WINDOWS / (bucket_ms * 1e-3) / 1e9, 1) if bucket_ms else None,
                        "frac": round(10.0 * N * MSM_WINDOWS / (bucket_ms * 1e-3) / 1e9 / MULPEAK_G, 3) if bucket_ms else None},
                "note": "integer-ALU bound (254-bit Montgomery arithmetic), see DESIGN.md; algorithmic bytes = 96 B x 2^20",
            },
            "extra": {
                "msm_only_ms": round(msm_ms, 4),
                "msm_only_mscalar_per_s": round(N / (msm_ms * 1e-3) / 1e6, 2) if msm_ms else None,
                "ntt_only_ms": round(ntt_ms, 4),
                "ntt_gfieldop_per_s": round(1.5 * N * K / (ntt_ms * 1e-3) / 1e9, 2) if ntt_ms else None,
                "ntt_algorithmic_GBps": round(64.0 * N / (ntt_ms * 1e-3) / 1e9, 1) if ntt_ms else None,
                "kernel_avg_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items() if v[1]},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(srs, column)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
