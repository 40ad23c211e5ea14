#!/usr/bin/env python3
"""VPD student train-step benchmark on MI355X (BASELINE.json metric: frame-crops/sec).

One "step" = one pass of the hot path over one resident synthetic batch:
  pack weights -> forward (train-mode BN) -> sum-MSE -> backward -> [RCCL SUM all-reduce] -> AdamW.
Workload (BASELINE.json configs[1]): ResNet-34 student, 5-channel 128x128 crops (RGB+flow),
emb_dim 128, batch 256 per GPU, bf16 operands / fp32 accumulation, weak scaling over GPUs.

  python bench.py --gpus N --steps K --warmup W
One rank per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a rank; started
from a bare shell with --gpus N > 1 it spawns the N ranks itself (before anything touches the GPU) and relays rank 0's
line.  W warm-up steps, then `--repeats` timed regions of EXACTLY K steps each, every one bracketed by barrier +
synchronize with the MAX over ranks; the reported value is the MEDIAN region, all of them are listed.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch

ARCH, C_IN, EMB_DIM, HW, BATCH_PER_GPU = "resnet34", 5, 128, 128, 256
# algorithmic work per crop: fwd + dgrad (no stem dgrad) + wgrad
TRAIN_FLOP_PER_CROP = {("resnet34", 5): 7203061760, ("resnet34", 3): 7100301312, ("resnet18", 5): 3579183104}
FWD_FLOP_PER_CROP = 2443837440               # ResNet-34, 5x128x128, D=128
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md
DIVING48_MEAN_STD = ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
                     (0.16302619019820488, 0.17092395707914718, 0.19266662199338647))


def train_flop_per_crop(arch, c_in, hw, emb_dim):
    """Algorithmic FLOP of one training crop: 2 * (3 * MACs_fwd - MACs_stem) -- forward, data gradient
    (none for the stem: the input needs no gradient) and weight gradient of every conv + fc."""
    layers = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3),
              "wide_resnet50_2": (3, 4, 6, 3), "wide_resnet101_2": (3, 4, 23, 3)}[arch]
    bott = arch not in ("resnet18", "resnet34")
    base = 128 if arch.startswith("wide") else 64
    h = (hw + 6 - 7) // 2 + 1
    stem = h * h * 64 * c_in * 49
    macs = stem
    h = (h + 2 - 3) // 2 + 1
    inpl = 64
    for s, (nblk, planes) in enumerate(zip(layers, (64, 128, 256, 512))):
        for b in range(nblk):
            stride = 2 if (b == 0 and s > 0) else 1
            ho = (h + 2 - 3) // stride + 1
            if bott:
                width, outc = planes * base // 64, planes * 4
                macs += h * h * width * inpl + ho * ho * width * width * 9 + ho * ho * outc * width
            else:
                outc = planes
                macs += ho * ho * planes * inpl * 9 + ho * ho * planes * planes * 9
            if stride != 1 or inpl != outc:
                macs += ho * ho * outc * inpl
            h, inpl = ho, outc
    macs += inpl * emb_dim
    return 2 * (3 * macs - stem)


def synthetic_batch(n, device, seed, c_in=C_IN, mean_std=DIVING48_MEAN_STD, target_dim=EMB_DIM):
    """Synthetic crops in the reference's value ranges (vpd_dataset/common.py:52-69): RGB ~ U{0..255}/255 normalised with
    the data set's mean/std; the other c_in - 3 channels are flow planes, clip(round(124+12 N(0,1)))/255 - 0.5; random
    teacher targets (2 * emb_dim wide with the motion head: train_vpd_model.py:61-65)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rgb = torch.randint(0, 256, (n, 3, HW, HW), generator=g, device=device).float() / 255.0
    mean = torch.tensor(mean_std[0], device=device).view(1, 3, 1, 1)
    std = torch.tensor(mean_std[1], device=device).view(1, 3, 1, 1)
    flow = (124 + 12 * torch.randn((n, c_in - 3, HW, HW), generator=g, device=device)).round().clamp(0, 255) / 255.0 - 0.5
    img = torch.cat([(rgb - mean) / std, flow], dim=1).contiguous()
    emb = torch.randn((n, target_dim), generator=g, device=device)
    return img, emb


# BASELINE.json configs run under their own names (--config); per-GPU batch so that N GPUs give the config's global batch
BENCH_CONFIGS = {
    "c2": dict(what="configs[1]: Diving48-shaped synthetic crops 128x128, %s student (5-ch RGB+flow), emb_dim 128, sum-MSE + AdamW",
               c_in=5, motion=False, norm="diving48", batch=256),
    "c3": dict(what="configs[2]: --motion two-stream (6-ch RGB+flow input), %s student, emb_dim 128 + motion head, sum-MSE + AdamW "
                    "(global batch 512 on 2 GPUs)", c_in=6, motion=True, norm="diving48", batch=256),
    "c4": dict(what="configs[3]: figure-skating-shaped crops 128x128 (fs normalisation), %s student (5-ch), emb_dim 128 + motion "
                    "head, sum-MSE + AdamW (global batch 4096 on 8 GPUs)", c_in=5, motion=True, norm="fs", batch=512),
}


def rccl_probe(eng, pl, world, device, iters=10):
    """Self-validation of the multi-GPU leg: what the communicator says it is, and the bus bandwidth of a SUM
    all-reduce of each gradient bucket's size, alone on the device (2 (N-1)/N x bytes / time: RCCL's convention)."""
    import torch.distributed as dist
    backend = dist.get_backend()
    info = {"backend": backend, "world_size": dist.get_world_size(), "rank0_device": str(device)}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    buckets = []
    for b, (off, numel) in enumerate(pl.buckets):
        if numel == 0:
            continue
        buf = torch.zeros(numel, dtype=torch.float32, device=device)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / iters
        buckets.append({"bucket": b, "MB": numel * 4 / 1e6, "ms": ms,
                        "busbw_GBps": 2.0 * (world - 1) / world * numel * 4 / (ms * 1e-3) / 1e9})
    info["allreduce_per_bucket"] = buckets
    return info


def host_cpu():
    """(threads to use, description): the PHYSICAL cores this process may be scheduled on -- capped by the container's cgroup
    CPU quota when there is one -- and the CPU model string.  /proc/cpuinfo gives (physical id, core id) per logical CPU;
    SMT siblings share a pair."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    model, cores, cur = "unknown CPU", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur and int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = [t.strip() for t in line.split(":", 1)]
            cur[k] = v
            if k == "model name":
                model = v
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    phys = len(cores) or len(allowed)
    # a container's CPU share is a cgroup bandwidth quota, not an affinity mask: more runnable threads than quota / period
    # are throttled (128 threads under a 16-CPU quota ran the oracle 5x slower than 16 threads)
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            q, per = parse(open(path).read())
            if per is None:
                per = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if q not in ("max", "-1"):
                quota = max(1, int(round(float(q) / float(per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    threads = min(phys, quota) if quota else phys
    desc = "%s, %d physical cores of %d schedulable logical CPUs" % (model, phys, len(allowed))
    if quota:
        desc += ", cgroup CPU quota %d" % quota
    return threads, desc


def cpu_baseline(batch=64, warm=3, timed=10):
    """BASELINE.md section 3 / SURVEY 8d: the CPU oracle (fp32 torch-CPU restatement of the reference loop, kind "port") on ALL
    physical host cores: ResNet-34, 5x128x128, D=128, --motion head on, B=64 synthetic crops; 3 warm-up + >= 10 timed train
    steps (fwd + sum-MSE + bwd + AdamW: train_vpd_model.py:67-98) and 3 + >= 10 timed embed() calls (models/rgb.py:72-86).
    Returns (train block, apply block)."""
    from oracle import vpd_oracle as O
    threads, desc = host_cpu()
    enc = O.reference_init_state_dict(ARCH, C_IN, EMB_DIM, 0)
    dec = O.procedural_state_dict(O.decoder_schema(EMB_DIM), 3)
    orc = O.StudentOracle(ARCH, C_IN, EMB_DIM, True, enc, dec)
    orc.get_optimizer(5e-4)
    img = O.synthetic_crops(batch, C_IN, HW, 1)
    tgt = O.synthetic_targets(batch, EMB_DIM, True, 2)
    # "all physical cores" is the plan, but a GPU box's CPUs are shared: with more threads than the
    # box really grants the oracle gets SLOWER (128 threads: 22.7 crops/s where 16 give ~5x that).  One probe step per
    # candidate count; the fastest is used for the timed legs and reported as `cores`.
    cand = sorted({t for t in (threads, 64, 32, 16, 8) if t <= threads}, reverse=True)
    probe = {}
    for t in cand:
        torch.set_num_threads(t)
        orc.train_step(img, tgt)
        t0 = time.perf_counter()
        orc.train_step(img, tgt)
        probe[t] = time.perf_counter() - t0
    threads = min(probe, key=probe.get)
    desc += "; threads picked by a one-step probe: " + ", ".join("%d: %.2f s" % (t, probe[t]) for t in cand)
    torch.set_num_threads(threads)
    for _ in range(warm):
        orc.train_step(img, tgt)
    t0 = time.perf_counter()
    for _ in range(timed):
        orc.train_step(img, tgt)
    dt = time.perf_counter() - t0
    train = {"value": batch * timed / dt, "unit": "crops/s", "cores": threads, "kind": "port",
             "sample": "%d warm-up + %d timed train steps of %d crops in %.1f s (ResNet-34, 5x128x128, D=128, motion head, "
                       "fp32, torch-CPU oracle; %s)" % (warm, timed, batch, dt, desc)}
    for _ in range(warm):
        O.embed(orc.enc, img, ARCH, True)
    t0 = time.perf_counter()
    for _ in range(timed):
        O.embed(orc.enc, img, ARCH, True)
    dt = time.perf_counter() - t0
    apply = {"value": batch * timed / dt, "unit": "crops/s", "cores": threads, "kind": "port",
             "sample": "%d warm-up + %d timed embed() calls of %d crops in %.1f s (same student, eval mode; %s)"
                       % (warm, timed, batch, dt, desc)}
    return train, apply


PARITY_FIXTURE = "tests/golden/c2_r34_c5_d128_m0_n256.npz"
PARITY_TOL = 2e-2


def parity_block(device):
    """The metric's second half, "emb L2 vs ref" (BASELINE.json): embed() (models/rgb.py:72-86) of the 256 crops of the committed
    fixture -- embeddings the REFERENCE produced on its CPU path for configs[1]'s student (oracle/gen_golden.py) -- through the HIP
    eval forward, per-sample ||e - e_ref|| / ||e_ref||.  Inputs (weights, crops) are regenerated from the fixture's seeds by
    tests/golden/recipe.py (numpy; nothing under oracle/ is imported).  Runs before the timed region, on its own model."""
    import importlib.util
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    spec = importlib.util.spec_from_file_location("vpd_golden_recipe", os.path.join(REPO, "tests", "golden", "recipe.py"))
    recipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(recipe)
    g = np.load(os.path.join(REPO, PARITY_FIXTURE))
    meta = json.loads(str(g["meta"]))
    enc = RGBF_EmbeddingModel(meta["arch"], meta["emb_dim"], meta["c_in"] != 3, device)
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in recipe.procedural_weights(shapes, meta["seed"]).items()})
    img = recipe.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1)
    e = np.asarray(enc.embed(img), np.float64)
    ref = np.asarray(g["emb_eval"], np.float64)
    ps = np.linalg.norm(e - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
    del enc
    return {"emb_rel_l2_max": float(ps.max()), "emb_rel_l2_mean": float(ps.mean()), "crops": int(meta["n"]), "fixture": PARITY_FIXTURE,
            "tol": PARITY_TOL, "ok": bool(ps.max() <= PARITY_TOL),
            "what": "per-sample ||e - e_ref|| / ||e_ref|| of embed() (eval mode, bf16 operands / fp32 accumulation) against the "
                    "reference's fp32 CPU embeddings of the same crops and weights (ResNet-34, 5x128x128, D=128)"}


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT touching the HIP runtime (the parent of the ranks must not initialise it):
    the visibility variables if set, else the KFD topology nodes that have SIMDs."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                props = dict(line.split() for line in open(os.path.join(root, node, "properties")) if len(line.split()) == 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return n


def spawn_ranks(n, argv):
    """--gpus N from a bare shell: start the N ranks as CHILD processes (this parent never initialises the GPU, and
    nothing is exec'ed from a process that has), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ngpu = visible_gpu_count()                        # (no torch.cuda.* call in the parent of the ranks)
    procs = []
    for r in range(n):
        # fewer devices than ranks (a 1-GPU box driving the 2-rank path over gloo): ranks share device r % ngpu
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r % max(ngpu, 1)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode(errors="replace")
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


def apply_block(enc, device, batches=30, warmup=4):
    """Inference twin (BASELINE configs[4]): 1,000-crop batches (500 frames x 2 views, apply_vpd_model.py:15) of the
    eval forward as ONE hipGraph launch per batch, inputs resident; the timed region is `batches` launches."""
    frames, k = 500, 2
    n = frames * k
    eng = enc.engine
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn((n, C_IN, HW, HW), generator=g, device=device)
    out = torch.empty((n, EMB_DIM), dtype=torch.float32, device=device)
    enc.eval()
    pl = eng.capture_eval_graph(x, out)
    for _ in range(warmup):
        eng.launch_eval_graph(pl, n)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(batches):
        eng.launch_eval_graph(pl, n)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    # the loop as apply_vpd_model.py runs it by default: decoded u8 frames from pinned host memory (82 KB per frame), views
    # [orig, h-flip] built on the device, hipGraph forward, D2H of the embeddings, per-video tuple lists
    from vpd_amd.apply import embed_dataset
    from vpd_amd.augment import CropAugmenter
    aug = CropAugmenter(device, DIVING48_MEAN_STD, HW, True)
    gc = torch.Generator().manual_seed(2)
    pool = [(torch.randint(0, 256, (frames, HW, HW, 3), generator=gc, dtype=torch.uint8).pin_memory(),
             torch.randint(100, 150, (frames, HW, HW, 2), generator=gc, dtype=torch.uint8).pin_memory()) for _ in range(2)]

    def loader(nb):
        for b in range(nb):
            idx = torch.arange(b * frames, (b + 1) * frames)
            yield {"video": idx // 157, "frame": idx % 157, "rgb_u8": pool[b % 2][0], "flow_u8": pool[b % 2][1]}
    embed_dataset(enc, loader(warmup), warmup * frames // 157 + 1, augmenter=aug, flip=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    embs = embed_dataset(enc, loader(batches), batches * frames // 157 + 1, augmenter=aug, flip=True)
    torch.cuda.synchronize(device)
    dt_u8 = time.perf_counter() - t0
    assert sum(len(v) for v in embs) == batches * frames
    enc.train()
    return {"tflops": batches * n / dt * FWD_FLOP_PER_CROP / 1e12,
            "loop_host_u8": {"value": batches * n / dt_u8, "unit": "crops/s",
                             "what": "vpd_amd.apply.embed_dataset on %d batches of %d u8 frames from pinned host memory: H2D, "
                                     "device-side views, hipGraph forward, D2H, tuple assembly (the 1 M-crop job with one pickle "
                                     "per video: profiles/r04_apply_bench_1M.json, tools/bench_apply.py --crops 1000000)"
                                     % (batches, frames)},
            "loop_vs_graph": dt / dt_u8,      # the whole loop's rate as a fraction of the bare graph launches' (same batches)
            "frac_of_mfma_peak": batches * n / dt * FWD_FLOP_PER_CROP / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
            "workload": "configs[4]-shaped: %d hipGraph launches of %d frames x %d views (1000 crops), eval forward, "
                        "inputs resident" % (batches, frames, k),
            "value": batches * n / dt, "unit": "crops/s", "ms_per_batch": 1e3 * dt / batches,
            "finite": bool(torch.isfinite(out).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps; the median is reported")
    ap.add_argument("--batch", type=int, default=None, help="crops per GPU per step (default: the config's)")
    ap.add_argument("--config", default="c2", choices=sorted(BENCH_CONFIGS),
                    help="BASELINE.json workload: c2 = configs[1] (default, the metric's config), c3 = configs[2] (6-channel input, "
                         "motion head, 256 crops per GPU), c4 = configs[3] (fs normalisation, motion head, 512 crops per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-apply", action="store_true", help="skip the inference-twin block")
    ap.add_argument("--no-parity", action="store_true", help="skip the embedding-parity block (emb L2 vs the reference fixture)")
    ap.add_argument("--arch", default=ARCH, help="student architecture (default: the BASELINE config, resnet34)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="element type of the HIP path: bf16 (BASELINE configs[1], default) or fp16 + static loss scale (the reference's own "
                         "GPU precision: fp16 autocast + GradScaler)")
    ap.add_argument("--profile-steps", type=int, default=3, help="event-instrumented steps after the timed region")
    args = ap.parse_args()

    cfg = BENCH_CONFIGS[args.config]
    if args.batch is None:
        args.batch = cfg["batch"]
    c_in, motion = cfg["c_in"], cfg["motion"]
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if os.environ.get("VPD_BENCH_STREAM") == "created":      # A/B: the step on a created stream instead of the null stream
        torch.cuda.set_stream(torch.cuda.Stream(device))
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VPD_DIST_BACKEND", "nccl")      # "nccl" is RCCL; gloo only for single-GPU dry runs of this path
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit("RCCL needs one GPU per rank (%d visible, %d ranks); VPD_DIST_BACKEND=gloo runs the ranks "
                             "on shared devices" % (torch.cuda.device_count(), world))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer

    from vpd_amd.data import RGB_MEAN_STD
    # "emb L2 vs ref": before anything is timed, on a model of its own (rank 0; every rank computes the same thing)
    parity = None
    if rank == 0 and not args.no_parity:
        parity = parity_block(device)
        if not parity["ok"]:
            raise SystemExit("bench.py: embeddings differ from the reference fixture: %s" % json.dumps(parity))
    torch.manual_seed(0)
    enc = RGBF_EmbeddingModel(args.arch, EMB_DIM, True, device, in_channels=c_in, dtype=args.dtype)
    enc.reset_parameters(seed=0)                 # reference init semantics, same weights on every rank
    trainer = ModelTrainer(enc, motion=motion)
    if world > 1 and motion:                     # the motion head is initialised in the trainer: rank 0's on every rank
        torch.distributed.broadcast(enc.engine.params, 0)
        enc.engine.mark_weights_changed()
    optimizer, scaler = trainer.get_optimizer(5e-4)
    img, emb = synthetic_batch(args.batch, device, seed=1 + rank, c_in=c_in, mean_std=RGB_MEAN_STD[cfg["norm"]],
                               target_dim=(2 if motion else 1) * EMB_DIM)
    eng = enc.engine
    enc.train()

    from vpd_amd.models.util import step

    def one_step():
        # the body of ModelTrainer.epoch for one batch (train_vpd_model.py:79-91): forward + loss, then step() =
        # loss.backward() [with the bucketed RCCL all-reduce when world > 1]; optimizer.step(); optimizer.zero_grad()
        loss = trainer._forward_loss(img, emb, train=True)
        step(optimizer, scaler, loss)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # one-time setup that is not a "step": build the plan / workspace, and bring the RCCL communicator up
    eng.plan(HW, HW, args.batch, True, motion)
    if world > 1:
        t = torch.zeros(1, device=device)
        torch.distributed.all_reduce(t)
    torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step()
    region_s = []
    for _ in range(max(args.repeats, 1)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        region_s.append(dt)
    dt = sorted(region_s)[len(region_s) // 2]            # median timed region of exactly --steps steps
    loss_now = float(eng.loss_step.item())

    # ---- instrumented steps for the roofline: EVERY rank runs them (they contain the gradient all-reduce; a rank that
    # ran them alone would wait for its peers forever), rank 0 arms the per-kernel dispatch events and reads them ----
    pl = eng.plan(HW, HW, args.batch, True, motion)
    if rank == 0:
        eng.set_timing(pl, True)
    for _ in range(args.profile_steps):
        one_step()
    sync()
    cls = {}
    if rank == 0:
        cls = eng.read_timing(pl) if args.profile_steps > 0 else {}
        eng.set_timing(pl, False)

    # multi-GPU self-validation (every rank takes part): the same K steps with the all-reduces NOT overlapped with backward
    # (VPD_DDP_OVERLAP=0: reduced in line after it), and the communicator's own numbers
    multi = None
    if world > 1:
        os.environ["VPD_DDP_OVERLAP"] = "0"
        for _ in range(3):
            one_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        sync()
        t_off = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t_off, op=torch.distributed.ReduceOp.MAX)
        os.environ.pop("VPD_DDP_OVERLAP", None)
        multi = rccl_probe(eng, pl, world, device)
        multi["ms_per_step_overlap_on"] = 1e3 * dt / args.steps
        multi["ms_per_step_overlap_off"] = 1e3 * float(t_off.item()) / args.steps
        multi["lazy_gradients"] = os.environ.get("VPD_DDP_LAZY", "1") != "0"
        multi["early_bucket0"] = bool(pl.early_bucket0)      # layer4's weight gradients at layer4's end: bucket 0 handed over early
        multi["wire_dtype"] = os.environ.get("VPD_DDP_WIRE", "fp32")      # bf16: gradient messages travel and are summed in bf16
        # replicas: every rank must hold the same bits after the timed steps (same initial weights, same all-reduced gradients)
        chk = torch.stack([eng.params.double().sum(), (eng.params.double() ** 2).sum()])
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        multi["replicas_identical"] = bool(torch.equal(lo, hi))

    # a grid-barrier time-out (fused BatchNorm backward; vpd_amd/csrc/sync.h) means the timed steps were not valid steps
    nerr = eng.sync_errors()
    if nerr:
        raise SystemExit("bench.py: %d in-launch barrier time-outs on rank %d: the measured steps are invalid" % (nerr, rank))

    out = None
    if rank == 0:
        crops = args.batch * world * args.steps
        value = crops / dt
        flop = train_flop_per_crop(args.arch, c_in, HW, EMB_DIM)
        assert (args.arch, c_in) not in TRAIN_FLOP_PER_CROP or flop == TRAIN_FLOP_PER_CROP[(args.arch, c_in)]
        if motion:      # FCNet(D, [128, 128], 2D) (models/module.py:133-156): forward + both gradients of three linear layers
            flop += 2 * 3 * (EMB_DIM * 128 + 128 * 128 + 128 * 2 * EMB_DIM)
        kernels = {}
        for k, v in cls.items():
            if v["launches"] > 0:
                kernels[k] = {"launches_per_step": v["launches"] / args.profile_steps,
                              "ms_per_step": v["ms"] / args.profile_steps,
                              "avg_launch_us": 1e3 * v["ms"] / v["launches"],
                              "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12}
        if not kernels:      # --profile-steps 0 (used under rocprofv3): no event timing, no roofline object
            kernels = {"(not timed)": {"ms_per_step": 0.0, "tflops": 0.0}}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        mat_ms = sum(v["ms"] for v in cls.values()) / max(args.profile_steps, 1)
        mat_flops = sum(v["flops"] for v in cls.values()) / max(args.profile_steps, 1)
        traffic, traffic_note, traffic_source = None, None, None
        try:      # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 note)
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
            # rows of the dominant class only: its kernel name AND one of its tile shapes ("<256,64,416>" -> "<256, 64, 416,")
            import re
            base = dom.split("<")[0].split(" ")[0]
            tiles = [", ".join(t.split(",")) for t in re.findall(r"<([0-9,]+)>", dom)]
            rows = [v for k, v in pmc["kernels"].items()
                    if base in k and (not tiles or any(("<" + t + ",") in k or ("<" + t + ">") in k for t in tiles))]
            n = sum(r["launches"] for r in rows)
            traffic = sum(r["launches"] * (2 * r["fetch_KB_per_launch"] + r["write_KB_per_launch"]) for r in rows) / n * 1024
            traffic_note = pmc["note"]
            from vpd_amd.boxid import gpu_unique_id
            traffic_source = dict(pmc.get("source") or {}, file="profiles/pmc_traffic.json")
            # the pool's containers share one hostname: the box is named by the GPU's unique_id (KFD topology)
            here = gpu_unique_id(local_rank)
            traffic_source["this_run_gpu_unique_id"] = here
            traffic_source["same_gpu_as_this_run"] = here != "unknown" and traffic_source.get("gpu_unique_id") == here
        except Exception:
            pass
        roofline = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"],
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": kernels[dom]["tflops"] / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_note": traffic_note, "traffic_source": traffic_source,
                    "whole_step_frac": value / world * flop / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12),
                    # FLOP-weighted over ALL matrix-kernel classes (= their FLOPs / their summed time); `frac` above
                    # is the single class that takes the most time
                    "matrix_kernels_frac": (mat_flops / (mat_ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS) if mat_ms > 0 else None,
                    "matrix_kernels_ms_per_step": mat_ms,
                    "kernels": kernels,
                    "note": "per-class HIP-event timing from %d instrumented steps run right after the timed region: the "
                            "start/stop events ride in each kernel's own dispatch packet (hipExtLaunchKernelGGL), so the "
                            "figure is the kernel's duration as rocprofv3 --kernel-trace reports it; a grouped weight-gradient "
                            "launch (all 3x3 stride-1 convs of one ResNet stage) is timed without its slab reduce" % args.profile_steps}
        out = {"metric": "frame-crops/sec (VPD student train)", "value": value, "unit": "crops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "repeats": {"n": len(region_s), "pick": "median", "crops_per_s": [args.batch * world * args.steps / t for t in region_s]},
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "gpu_unique_id": __import__("vpd_amd.boxid", fromlist=["gpu_unique_id"]).gpu_unique_id(local_rank),
               "config": {"workload": cfg["what"] % ("ResNet-34" if args.arch == ARCH else args.arch) + ", batch=%d per GPU" % args.batch,
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "flop_per_crop": flop, "loss_last_step": loss_now},
               "roofline": roofline}
        if parity is not None:
            out["parity"] = parity
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_apply and args.arch == ARCH and args.config == "c2":
            out["apply"] = apply_block(enc, device)
        if world == 1 and not args.no_cpu_baseline and args.arch == ARCH and args.config == "c2":
            out["cpu_baseline"], apply_cpu = cpu_baseline()
            if "apply" in out:
                out["apply"]["cpu_baseline"] = apply_cpu
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
