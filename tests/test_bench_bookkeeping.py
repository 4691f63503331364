"""bench.py's rank / WORLD_SIZE bookkeeping on CPU (gloo, world size 2): every rank times the same K steps between
barriers, the reported time is the MAX over ranks and `value` is the graphs ALL ranks sampled divided by it (the
driver's contract for N > 1).  Plus the shape of the `roofline` / `window_attn` objects from synthetic profiler tallies."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

STEPS, WARMUP, BATCH = 3, 1, 7
SLEEP = (0.02, 0.10)   # per-step wall time of rank 0 / rank 1: rank 1 is the slow one


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import bench
    assert bench.rank_env() == (rank, rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(time.perf_counter())
        time.sleep(SLEEP[rank])

    per_rank = []
    dt = bench.timed_steps(step, STEPS, WARMUP, lambda: None, dist, None, per_rank=per_rank)
    # every rank's own wall time, in rank order, on every rank: the slow rank's is the maximum (and <= the reported one)
    assert len(per_rank) == world and per_rank[1] > per_rank[0] and max(per_rank) <= dt
    assert abs(per_rank[rank] - STEPS * SLEEP[rank]) < 0.3
    q.put((rank, dt, len(calls), bench.whole_job_value(BATCH, world, STEPS, dt)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_is_max_over_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, dt0, n0, v0), (_, dt1, n1, v1) = got
    assert n0 == n1 == STEPS + WARMUP                      # exactly K timed steps after W warm-ups, on every rank
    assert dt0 == dt1                                      # MAX over ranks, the same number everywhere
    assert STEPS * SLEEP[1] <= dt0 < STEPS * SLEEP[1] + 0.5  # the slow rank sets it
    assert v0 == v1 == BATCH * 2 * STEPS / dt0             # whole-job graphs / max wall


def test_single_process_path_and_json_objects():
    import bench
    n = []
    own = []
    dt = bench.timed_steps(lambda: n.append(1), 4, 2, lambda: None, per_rank=own)
    assert len(n) == 6 and dt >= 0.0 and len(own) == 1 and own[0] <= dt
    assert bench.cpu_thread_counts()[-1] >= 1
    assert bench.whole_job_value(112, 8, 2, 4.0) == 448.0
    kern = {
        "k_mlp_stream<384>": dict(ms=12.0, flops=6 * 152e9, bytes=6 * 0.2e9, launches=6),
        "k_wattn_res": dict(ms=6.0, flops=2 * 90e9, bytes=2 * 0.79e9, launches=2),
        "k_wattn_ring<192>": dict(ms=6.6, flops=2 * 83e9, bytes=2 * 0.4e9, launches=2),
        "k_wattn_ring<384>": dict(ms=7.0, flops=6 * 60e9, bytes=6 * 0.1e9, launches=6),
        "k_wattn_core": dict(ms=0.9, flops=2 * 1.8e9, bytes=2 * 0.1e9, launches=2),
        "k_row<cond_ln>": dict(ms=4.0, flops=0.0, bytes=12 * 0.15e9, launches=12),
    }
    roof, wattn = bench.roofline_objects(kern, 112, "none")
    assert roof["kernel"] == "k_mlp_stream<384>" and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    assert abs(roof["achieved"] - 152e9 / 2e-3 / 1e12) < 1e-6 and abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-12
    assert roof["traffic"] is None and set(roof["kernels"]) == set(kern)
    assert wattn["s0_C96"]["kernel"] == "k_wattn_res" and abs(wattn["s0_C96"]["tflops"] - 30.0) < 0.1
    assert abs(wattn["s0_C96"]["mfma_util_pct"] - 1.2) < 0.01
    total = (2 * 90e9 + 2 * 83e9 + 6 * 60e9) / ((6.0 + 6.6 + 7.0) * 1e-3) / 1e12
    assert abs(wattn["fused_kernels_mfma_util_pct"] - 100 * total / 2500.0) < 0.01
