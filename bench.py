#!/usr/bin/env python3
"""Headline benchmark: Point2Net backbone forward, points/s at N=8192, B=16 per GPU
(BASELINE.json `metric`, configs[1]), scene-sharded over the GPUs of one node.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one eval forward of the drop-in module (`src.models.point2net.Point2Net`,
trained weights from tests/golden/weights.npz) over one resident synthetic
ScanNet-shaped batch of 16 x 8192 points.  Every rank owns its own batch and there is
no data-path collective (eval clouds are independent: SURVEY.md section 8e), so
scaling is weak and `value` = points all ranks processed / max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      for the dominant kernel, from HIP events recorded around every kernel
                launch inside the timed region (libp2n's p2n_profile_* hooks)
  cpu_baseline  the drop-in module on CPU tensors: stock Conv1d/BatchNorm1d/ReLU, the reference's own op
                sequence (its Python files do not travel to the GPU box), timed on this host
  cpu_port      the CPU oracle (oracle/p2n_oracle.c, double accumulation) timed on this host
  fps           farthest-point sampling of one 40k-point scene (the metric's second half)
  train         one training step (BASELINE.json configs[3]: train-mode forward + InstanceSegLoss + backward + clip + Adam)
                at B=32 x 8192 on this GPU, ms per step in fp32 and in the opt-in bf16 mode (tools/bench_train.py)
  hier          the hierarchical sibling (FPS + ball query + shared MLPs + three-NN propagation) at the
                same B=16 x N=8192: ms per forward and the shared-MLP kernels against the fp32 MFMA peak
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

B_PER_GPU = 16
N_POINTS = 8192
# algorithmic FLOPs per point ("Roofline"): 2 * MACs of the layers a
# kernel runs, with the global half of head layer 6 folded to a per-cloud bias
FLOP_PER_PT = {"encoder": 2 * 127360, "head": 2 * 106496}
FLOP_PER_PT_STEP = 467712          # SURVEY.md section 8d (598,784 as the reference writes it)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
KERNEL_IDS = {"encoder": 0, "mid": 1, "head": 2, "fps": 3}
DOM_KERNEL = "encoder"              # the longest kernel of the step: the one the roofline object prices
DOM_KERNEL_ID = KERNEL_IDS[DOM_KERNEL]
PREWARM_STEPS = 150


def load_model(device):
    from src.models.point2net import Point2Net
    sd = np.load(os.path.join(REPO, "tests", "golden", "weights.npz"))
    m = Point2Net(feature_dim=6, embedding_dim=64)
    m.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files}, strict=True)
    return m.to(device).eval()


def collect_kernel_ms():
    from pointcloudsegmentation_amd import _lib
    ms = (ctypes.c_float * 32)()
    cnt = (ctypes.c_int * 32)()
    _lib.check(_lib.lib().p2n_profile_collect(32, ms, cnt), "p2n_profile_collect")
    return {k: (ms[i] / cnt[i] if cnt[i] else None, cnt[i]) for k, i in KERNEL_IDS.items()}


def cpu_baselines(x_np):
    """Bounded CPU legs on rank 0: the oracle port (all OpenMP threads) and the drop-in
    module on stock PyTorch CPU ops.  Sample = the full 16 x 8192 batch, once."""
    from oracle import oracle
    from pointcloudsegmentation_amd.params import flatten_numpy
    from src.models.point2net import Point2Net
    sd = np.load(os.path.join(REPO, "tests", "golden", "weights.npz"))
    params = flatten_numpy({k: sd[k] for k in sd.files})
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    oracle.forward_eval(x_np[:1, :256], params)          # warm (loads the .so, spins up OpenMP)
    t = time.perf_counter()
    oracle.forward_eval(x_np, params)
    dt = time.perf_counter() - t
    port = {"value": x_np.shape[0] * x_np.shape[1] / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"{x_np.shape[0]}x{x_np.shape[1]} points (one full step), oracle/p2n_oracle.c, "
                      f"OpenMP over points, fp32 with double accumulation, {dt:.2f} s"}
    torch.set_num_threads(threads)
    m = Point2Net(6, 64)
    m.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files})
    m.eval()
    xt = torch.from_numpy(x_np)
    with torch.no_grad():
        m(xt[:2])
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            m(xt)
            ts.append(time.perf_counter() - t)
    dt = sorted(ts)[1]
    tcpu = {"value": x_np.shape[0] * x_np.shape[1] / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"drop-in module on CPU tensors (stock Conv1d/BatchNorm1d/ReLU = the reference's op "
                      f"sequence, reference point2net.py:36-48,67-87), median of 3 full steps of "
                      f"{x_np.shape[0]}x{x_np.shape[1]} points, {dt:.3f} s"}
    return port, tcpu


def fps_leg(device):
    """FPS of one 40,000-point scene to 10,000 samples: per-call ms (HIP events)."""
    from pointcloudsegmentation_amd import ops
    from pointcloudsegmentation_amd.synth import make_batch
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(1, 40000, 7)[:, :, :3])).to(device)
    npoint = 10000
    ops.farthest_point_sample(xyz, npoint)
    torch.cuda.synchronize(device)
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.farthest_point_sample(xyz, npoint)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    # eight scenes in one call: the per-iteration exchange latency is shared by the clouds in flight
    xyz8 = torch.from_numpy(np.ascontiguousarray(make_batch(8, 40000, 9)[:, :, :3])).to(device)
    ops.farthest_point_sample(xyz8, npoint)
    torch.cuda.synchronize(device)
    e0.record()
    ops.farthest_point_sample(xyz8, npoint)
    e1.record()
    torch.cuda.synchronize(device)
    ms8 = e0.elapsed_time(e1)
    def once(B, N, M, seed, kind="scannet"):
        pts = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, seed, kind)[:, :, :3])).to(device)
        ops.farthest_point_sample(pts, M)
        torch.cuda.synchronize(device)
        e0.record()
        ops.farthest_point_sample(pts, M)
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)
    # Floor of the round design at this shape: per accepted sample the update's instruction issue on
    # one SIMD (10 points per lane = 5 point pairs x 10 packed instructions x 4 cycles) plus one exchange per round -- one
    # producer->consumer hop inside an XCD, 530 shader cycles (profiles/r03_g_micro_fps_handoff.txt: 221-254 ns) -- spread
    # over the 8.6 samples a round accepts here (profiles/r03_h_fps_phase_clocks.txt), at the 2.35 GHz the kernel runs at.
    floor_us = (5 * 10 * 4 + 530 / 8.6) / 2350.0
    return {"ms_per_40k_scene": ms, "npoint": npoint, "us_per_sample": 1e3 * ms / npoint,
            "floor_us_per_sample": floor_us, "achieved_fraction_of_floor": floor_us / (1e3 * ms / npoint),
            "floor_model": "update issue (5 point pairs x 10 packed instructions x 4 cycles per sample) + one in-XCD hand-off "
                           "(530 cycles, profiles/r03_g_micro_fps_handoff.txt) per round of 8.6 samples "
                           "(profiles/r03_h_fps_phase_clocks.txt) at 2.35 GHz; publish, poll skew and the decision are what "
                           "the achieved figure adds",
            "ms_per_40k_scene_batch_of_8": ms8 / 8,
            "ms_16x8192_to_2048": once(16, 8192, 2048, 11),             # the hierarchical sibling's first level
            "ms_1m_to_16384": once(1, 1_000_000, 16384, 4, "lidar"),    # BASELINE.json configs[4]
            "members_per_cloud": {"8192": ops._lib.lib().p2n_fps_workgroups_per_cloud(16, 8192),
                                  "40000": ops._lib.lib().p2n_fps_workgroups_per_cloud(1, 40000),
                                  "1000000": ops._lib.lib().p2n_fps_workgroups_per_cloud(1, 1000000)},
            "note": "cluster kernels accept several samples per exchange round (every published best that provably is the "
                    "sequential algorithm's next sample); indices bit-identical to the one-at-a-time algorithm",
            "status": ops.fps_status(device)}


def bf16_leg(model, x, lib, steps):
    """The same step on the opt-in bf16 MFMA path (north_star: 'bf16/fp32 MFMA GEMMs', 2e-2 parity):
    reported beside the headline, never as it -- the reference computes in fp32."""
    with torch.no_grad():
        ref = model(x)
        model.precision = "bf16"
        try:
            for _ in range(3):
                y = model(x)
            torch.cuda.synchronize()
            lib.p2n_profile_enable(1)
            t0 = time.perf_counter()
            for _ in range(steps):
                y = model(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            k = collect_kernel_ms()
            lib.p2n_profile_enable(0)
        finally:
            model.precision = "fp32"
    pts = x.shape[0] * x.shape[1]
    dev = float((y - ref).abs().max() / ref.abs().max())
    # algorithmic HBM bytes per point: x 24 + bf16 f3 image 512 written + 512 read + emb 256
    bytes_pt = 24 + 512 + 512 + 256
    return {"value": pts * steps / dt, "unit": "points/s", "ms_per_step": 1e3 * dt / steps, "dtype": "bf16 (fp32 accumulate)",
            "kernels_ms": {n: v[0] for n, v in k.items() if v[1]}, "max_dev_vs_fp32_path": dev,
            "hbm_algorithmic_GB_per_s": bytes_pt * pts * steps / dt / 1e9,
            "note": "opt-in path (model.precision = 'bf16'); layers 0-2 split bf16, layers 3-8 bf16, epilogues fp32"}


def x3_leg(model, x, lib, steps):
    """The same step with fp32-accurate products on the bf16 matrix pipe (model.precision = 'fp32x3': three-way bf16
    splits, six exact part products, fp32 accumulate; same 1e-5 parity bar as the headline, tests/test_forward_x3_gpu.py)."""
    with torch.no_grad():
        ref = model(x)
        model.precision = "fp32x3"
        try:
            for _ in range(50):
                y = model(x)
            torch.cuda.synchronize()
            # as in the headline: one kernel bracketed by HIP events at a time (bracketing all three costs ~20 us of a step)
            k = {}
            for name in ("head", "mid", "encoder"):
                lib.p2n_profile_select(KERNEL_IDS[name])
                lib.p2n_profile_enable(1)
                t0 = time.perf_counter()
                for _ in range(steps):
                    y = model(x)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0          # the last pass (encoder bracketed, like the headline's timed region) is the one reported
                k[name] = collect_kernel_ms()[name]
                lib.p2n_profile_enable(0)
            lib.p2n_profile_select(-1)
        finally:
            model.precision = "fp32"
    pts = x.shape[0] * x.shape[1]
    dev = float((y - ref).abs().max() / ref.abs().max())
    enc_ms, head_ms = k["encoder"][0], k["head"][0]
    return {"value": pts * steps / dt, "unit": "points/s", "ms_per_step": 1e3 * dt / steps,
            "dtype": "f32 in/out, 3xbf16 split products (6 of 9 part products), fp32 accumulate",
            "kernels_ms": {n: v[0] for n, v in k.items() if v[1]}, "max_dev_vs_exact_fp32_path": dev,
            "encoder_speedup_vs_exact_fp32": None,
            "encoder_tflops_algorithmic": FLOP_PER_PT["encoder"] * pts / (enc_ms * 1e-3) / 1e12,
            "encoder_matrix_pipe_floor_ms": 6 * FLOP_PER_PT["encoder"] * pts / 2.5e15 * 1e3,
            "head_tflops_algorithmic": FLOP_PER_PT["head"] * pts / (head_ms * 1e-3) / 1e12,
            "note": "opt-in (model.precision = 'fp32x3'); the headline stays the exact-fp32 MFMA path.  "
                    "matrix_pipe_floor = 6 bf16 MFMA products per algorithmic MAC at the dense bf16 peak (2.5 PFLOP/s)"}


def train_leg(device, steps=10):
    """BASELINE.json configs[3] on one GPU: ms per training step at B=32 x 8192, fp32 and the opt-in bf16 mode."""
    from pointcloudsegmentation_amd.synth import make_batch, make_labels
    from src.models.losses import InstanceSegLoss
    from src.models.point2net import Point2Net
    B, N = 32, N_POINTS
    x = torch.from_numpy(make_batch(B, N, seed=200)).to(device)
    y = torch.from_numpy(make_labels(B, N, seed=0)).to(device)
    pts = x[:, :, :3].contiguous()
    out = {"workload": "train fwd + InstanceSegLoss + bwd + clip_grad_norm_ + Adam (reference train.py:118-136), "
                       "B=32 x 8192, random-init weights, synthetic labels", "steps": steps}
    for prec in ("fp32", "fp32x3", "bf16"):
        torch.manual_seed(0)
        model = Point2Net(6, 64).to(device).train()
        model.precision = prec
        crit = InstanceSegLoss(64, 0.1, 0.5)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)

        def step():
            opt.zero_grad()
            crit.prepare(pts, y)
            loss = crit(pts, model(x), y)["total_loss"]
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            opt.step()
            return loss

        for _ in range(3):
            step()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize(device)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        assert bool(torch.isfinite(loss))
        out[prec] = {"ms_per_step": ms, "points_per_s": B * N / (ms * 1e-3),
                     "tflops_algorithmic": 3 * FLOP_PER_PT_STEP * B * N / (ms * 1e-3) / 1e12,
                     "frac_of_fp32_mfma_peak": 3 * FLOP_PER_PT_STEP * B * N / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        del model, opt
    del out["bf16"]["frac_of_fp32_mfma_peak"]   # HBM-bound in this mode: the fp32 MFMA peak is not its roofline
    del out["fp32x3"]["frac_of_fp32_mfma_peak"]   # runs on the bf16 matrix pipe: six part products per algorithmic MAC
    out["fp32x3"]["dtype"] = "f32 in/out, 3xbf16 split products (6 of 9 part products), fp32 accumulate"
    out["fp32x3"]["note"] = ("opt-in (model.precision = 'fp32x3'): same reference goldens at the same tolerances as fp32 "
                             "(tests/test_train_x3_gpu.py); first layer (K = 6) on the fp32 MFMA")
    out["bf16"]["note"] = "GEMM operands rounded to bf16, fp32 accumulation and storage"
    torch.cuda.empty_cache()
    return out


def _note(msg):
    """progress on stderr (stdout carries the one JSON line): a watcher sees which leg a long run is in"""
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def _coll_device(dist, device):
    """where the bench's own small reductions live: the GPU under RCCL, the host under gloo (rehearsal)"""
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def _max_over_ranks(dist, device, seconds):
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _fence(dist, device):
    torch.cuda.synchronize(device)
    dist.barrier()
    torch.cuda.synchronize(device)


def train_leg_dist(dist, device, rank, world, steps=10):
    """BASELINE.json configs[3] at N > 1: the global batch of 32 x 8192 split over the ranks, parallel.train_step
    (forward + InstanceSegLoss + backward + ONE flat gradient all-reduce over RCCL + clip + Adam) inside the timed
    region; the all-reduce's own time from HIP events around a stand-alone repeat on the same bucket."""
    from pointcloudsegmentation_amd import parallel
    from pointcloudsegmentation_amd.synth import make_batch, make_labels
    from src.models.losses import InstanceSegLoss
    from src.models.point2net import Point2Net
    B, N = 32, N_POINTS
    assert B % world == 0, "the global batch of 32 clouds is split evenly over 1/2/4/8 ranks"
    b = B // world
    x = torch.from_numpy(make_batch(b, N, seed=200 + rank)).to(device)
    y = torch.from_numpy(make_labels(b, N, seed=rank)).to(device)
    out = {"workload": "train fwd + InstanceSegLoss + bwd + gradient all-reduce + clip_grad_norm_ + Adam (reference "
                       "train.py:118-136), global B=32 x 8192 split over the ranks, random-init weights, synthetic labels",
           "steps": steps, "clouds_per_rank": b, "scaling": "strong", "backend": dist.get_backend()}
    for prec in ("fp32", "fp32x3"):
        torch.manual_seed(0)                      # same initial weights on every rank
        model = Point2Net(6, 64).to(device).train()
        model.precision = prec
        crit = InstanceSegLoss(64, 0.1, 0.5)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
        for _ in range(3):
            loss = parallel.train_step(model, crit, opt, x, y)["total_loss"]
        _note(f"train {prec}: warm")
        _fence(dist, device)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = parallel.train_step(model, crit, opt, x, y)["total_loss"]
        _fence(dist, device)
        ms = 1e3 * _max_over_ranks(dist, device, time.perf_counter() - t0) / steps
        _note(f"train {prec}: {ms:.3f} ms per step")
        assert bool(torch.isfinite(loss))
        # the collective alone, on the bucket the backward left behind (1.21 MB, in place)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        parallel.allreduce_gradients(model.parameters())
        _fence(dist, device)
        e0.record()
        for _ in range(10):
            parallel.allreduce_gradients(model.parameters())
        e1.record()
        torch.cuda.synchronize(device)
        ar_ms = _max_over_ranks(dist, device, e0.elapsed_time(e1) / 10 * 1e-3) * 1e3
        out[prec] = {"ms_per_step": ms, "points_per_s": B * N / (ms * 1e-3), "allreduce_ms": ar_ms,
                     "allreduce_bytes": 4 * sum(p.numel() for p in model.parameters())}
        del model, opt
    torch.cuda.empty_cache()
    return out


def scenes_leg_dist(dist, device, rank, world, model):
    """BASELINE.json configs[2]: full scenes of 40k / 80k / 150k points, host arrays in -> host labels and embeddings out
    (normalise -> forward -> cluster, inference.predict_many), the scene list dealt round-robin over the ranks
    (sharding.shard_indices), no collective on the data path."""
    from pointcloudsegmentation_amd.inference import predict_many
    from pointcloudsegmentation_amd.sharding import shard_indices
    sizes = (40000, 80000, 150000)
    per_rank = 6
    n_scenes = per_rank * world
    mine = shard_indices(n_scenes, rank, world)

    def scene(i):
        r = np.random.default_rng(500 + i)
        n = sizes[(i // world) % len(sizes)]      # every rank gets two scenes of each size
        return ((r.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64),
                r.integers(0, 256, (n, 3)).astype(np.float64))
    sc = [scene(i) for i in mine]
    list(predict_many(model, sc[:3]))                 # allocators and pinned rings reach their steady state
    _fence(dist, device)
    t0 = time.perf_counter()
    res = list(predict_many(model, sc))
    torch.cuda.synchronize(device)
    dt = _max_over_ranks(dist, device, time.perf_counter() - t0)
    pts = sum(len(p) for p, _ in sc)
    tot = torch.tensor([float(pts)], dtype=torch.float64, device=_coll_device(dist, device))
    dist.all_reduce(tot)
    assert all(r["instance_labels"].shape[0] == len(p) for r, (p, _) in zip(res, sc))
    return {"workload": "scene stream, 40k/80k/150k-point scenes, normalise + forward + cluster, host in -> host out "
                        "(reference inference.py:143-188), scenes dealt round-robin over the ranks",
            "scenes": n_scenes, "scenes_per_rank": per_rank, "seconds": dt, "scenes_per_s": n_scenes / dt,
            "points_per_s": float(tot.item()) / dt, "scaling": "weak"}


def lidar_leg_dist(dist, device, rank, world):
    """BASELINE.json configs[4]: one 1M-point LiDAR-style cloud per rank: FPS to 16,384 samples, then kNN k=32 of the
    samples in the cloud; scene-parallel (replicas, no collective)."""
    from pointcloudsegmentation_amd import ops
    from pointcloudsegmentation_amd.synth import make_batch
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(1, 1_000_000, seed=40 + rank, kind="lidar")[:, :, :3])).to(device)
    M, k = 16384, 32

    def run():
        idx = ops.farthest_point_sample(xyz, M)
        q = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        return idx, ops.knn(q, xyz, k)[0]
    run()
    _fence(dist, device)
    t0 = time.perf_counter()
    idx, nn = run()
    torch.cuda.synchronize(device)
    dt = _max_over_ranks(dist, device, time.perf_counter() - t0)
    ops.fps_check(device)
    assert nn.shape == (1, M, k)
    return {"workload": "1M-point cloud per rank: FPS -> 16,384 + kNN k=32 (BASELINE.json configs[4]), scene-parallel",
            "ms_per_cloud": 1e3 * dt, "clouds_per_s": world / dt, "scaling": "weak"}


def pmc_traffic(kernel):
    """(HBM bytes per launch, provenance) of this workload's kernel from the committed PMC summary (counters
    cannot be read from inside the process); (None, None) when the file is absent.  The provenance names the
    commit whose kernels were measured, so a stale figure shows."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d["bytes_per_launch"].get(kernel), f"{d.get('source', '?')}; kernels as of commit {d.get('commit', '?')}"
    except (OSError, KeyError, ValueError):
        return None, None


def scene_leg(device, model, steps=5):
    """BASELINE.json configs[0]: one reference-sized scene (166,499 points) through inference.predict -- normalise ->
    forward -> embedding clustering (reference inference.py:76-188) -- and the clustering alone on two kinds of rows."""
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    from pointcloudsegmentation_amd.inference import predict
    from pointcloudsegmentation_amd.synth import make_embeddings
    n = 166_499
    rng = np.random.default_rng(0)
    pts = (rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
    col = rng.integers(0, 256, (n, 3)).astype(np.float64)

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize(device)
        return 1e3 * (time.perf_counter() - t0) / steps, r
    out = {"workload": "one 166,499-point scene, normalise + forward + DBSCAN(eps 0.1, min_samples 50) + relabel", "n": n}
    out["ms_host_arrays_in_host_labels_out"], res = timed(lambda: predict(model, pts, col))
    pg, cg = torch.from_numpy(pts).to(device), torch.from_numpy(col).to(device)
    out["ms_device_resident"], _ = timed(lambda: predict(model, pg, cg, to_host=False))
    out["num_instances"] = int(res["num_instances"])
    for order in ("random", "runs"):
        e = torch.from_numpy(make_embeddings(n, 64, 11, order=order)).to(device)
        out[f"cluster_ms_33_instances_rows_{order}"], lab = timed(lambda: cluster_embeddings(e))
    out["clusters"] = int(lab.max())
    out["note"] = ("the clustering works on the rows in the order of a code of their embeddings, so the caller's row order "
                   "does not matter (DESIGN.md section 10); labels bit-identical to scikit-learn's on the golden fixtures")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-fps", action="store_true", help="skip the FPS leg")
    ap.add_argument("--no-hier", action="store_true", help="skip the hierarchical-sibling leg")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg")
    ap.add_argument("--no-scenes", action="store_true", help="skip the scene legs (one scene at N = 1, the stream at N > 1)")
    args = ap.parse_args()

    if os.environ.get("P2N_BENCH_WATCHDOG"):      # debugging aid: dump every thread's Python stack if still running after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["P2N_BENCH_WATCHDOG"]), exit=False, file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no fallback)")
    # P2N_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- the N>1 control flow (rendezvous, barriers, MAX over ranks,
    # rank-0 line) on a one-GPU box; its number is NOT a scaling measurement and the line says so in `config`.
    rehearse = world > 1 and os.environ.get("P2N_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from pointcloudsegmentation_amd import _lib
    from pointcloudsegmentation_amd.synth import make_batch

    model = load_model(device)
    x_np = make_batch(B_PER_GPU, N_POINTS, seed=1000 + rank)      # every rank: its own scenes
    x = torch.from_numpy(x_np).to(device)
    lib = _lib.lib()

    with torch.no_grad():
        # clock ramp: the first ~100 steps (50 ms) after a cold start run 5-10 % slower whatever W is, so a
        # fixed, untimed pre-warm precedes the W warm-up steps (reported as config.prewarm_steps)
        for _ in range(PREWARM_STEPS):
            model(x)
        for _ in range(args.warmup):
            y = model(x)
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        # The roofline kernel (the encoder: the larger of the two MFMA kernels) is bracketed by HIP events
        # inside the timed region; bracketing every kernel costs ~20 us of a 0.5 ms step, so the other
        # kernels' durations come from a few extra, untimed steps below.
        lib.p2n_profile_select(DOM_KERNEL_ID)
        lib.p2n_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x)
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        dom_timed = collect_kernel_ms()
        lib.p2n_profile_select(-1)
        lib.p2n_profile_enable(1)
        for _ in range(10):
            model(x)
        torch.cuda.synchronize(device)
    kernels = collect_kernel_ms()
    lib.p2n_profile_enable(0)
    kernels[DOM_KERNEL] = dom_timed[DOM_KERNEL]       # the timed region's own measurement
    elapsed = t1 - t0
    if dist is not None:
        elapsed = _max_over_ranks(dist, device, elapsed)
    assert y.shape == (B_PER_GPU, N_POINTS, 64) and bool(torch.isfinite(y).all())

    # N > 1: the configurations a scaling run is meant to measure besides the headline, every rank taking part
    # (max-over-ranks timing inside each leg); the headline line's keys stay as they are
    dist_legs = {}
    if dist is not None:
        _note("headline done")
        if not args.no_train:
            dist_legs["train"] = train_leg_dist(dist, device, rank, world)
            _note("train leg done")
        if not args.no_scenes:
            dist_legs["scenes"] = scenes_leg_dist(dist, device, rank, world, model)
            _note("scenes leg done")
        if not args.no_fps:
            dist_legs["lidar"] = lidar_leg_dist(dist, device, rank, world)
            _note("lidar leg done")

    if rank == 0:
        pts_step = B_PER_GPU * N_POINTS
        value = world * pts_step * args.steps / elapsed
        dom = DOM_KERNEL
        dom_ms = kernels[dom][0]
        assert dom_ms >= (kernels["head"][0] or 0.0), "the encoder is expected to be the longest kernel of the step"
        achieved = FLOP_PER_PT[dom] * pts_step / (dom_ms * 1e-3) / 1e12
        out = {
            "metric": "backbone fwd points/sec (N=8192,B=16)",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Point2Net eval forward, B=16 x N=8192 ScanNet-shaped clouds per GPU "
                                   "(BASELINE.json configs[1]), trained weights, scene-sharded, no collective",
                       "global_batch": world * B_PER_GPU, "points_per_cloud": N_POINTS, "parallelism": f"scenes x{world}",
                       "prewarm_steps": PREWARM_STEPS,
                       **({"rehearsal": "all ranks on cuda:0 over gloo: control flow only, not a scaling figure"} if rehearse else {})},
            "per_gpu_value": value / world,
            "step_tflops_algorithmic": FLOP_PER_PT_STEP * pts_step / (elapsed / args.steps) / 1e12,
            "kernels_ms": {k: v[0] for k, v in kernels.items() if v[1]},
            "roofline": {"bound": "mfma", "kernel": f"p2n_{dom}_kernel", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": pmc_traffic(dom)[0], "traffic_source": pmc_traffic(dom)[1],
                         "note": f"{FLOP_PER_PT[dom]} algorithmic FLOP/pt x {pts_step} pts per launch / "
                                 f"{dom_ms:.4f} ms mean launch (HIP events on the launch stream, {kernels[dom][1]} launches); "
                                 "traffic = HBM bytes per launch from the committed rocprofv3 PMC passes "
                                 "(profiles/pmc_traffic.json), not collected live"},
        }
        out.update(dist_legs)
        if world == 1:
            out["bf16"] = bf16_leg(model, x, lib, args.steps)
            out["fp32x3"] = x3_leg(model, x, lib, args.steps)
            out["fp32x3"]["encoder_speedup_vs_exact_fp32"] = kernels["encoder"][0] / out["fp32x3"]["kernels_ms"]["encoder"]
            if not args.no_fps:
                out["fps"] = fps_leg(device)
            if not args.no_train:
                out["train"] = train_leg(device)
            if not args.no_hier:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_hier
                out["hier"] = bench_hier.run(B_PER_GPU, N_POINTS, iters=20)
                # the shared-MLP kernels alone on the chip (geometry in order on the same stream): their own fraction of
                # the fp32 MFMA peak, beside the figure above where FPS / ball query / three-NN co-run on side streams
                ser = bench_hier.run(B_PER_GPU, N_POINTS, iters=20, serial=True)
                out["hier"]["mlp_without_corunning_geometry"] = ser["mlp"]
                out["hier"]["ms_per_forward_wall_one_stream"] = ser["ms_per_forward_wall"]
            if not args.no_scenes:
                out["scene"] = scene_leg(device, model)
            if not args.no_cpu:
                port, tcpu = cpu_baselines(x_np)
                out["cpu_baseline"] = tcpu
                out["cpu_port"] = port
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
